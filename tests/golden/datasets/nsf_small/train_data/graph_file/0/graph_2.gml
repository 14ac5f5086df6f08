graph [
  directed 1
  node [
    id 0
    label "0"
  ]
  node [
    id 1
    label "1"
  ]
  node [
    id 2
    label "2"
  ]
  node [
    id 3
    label "3"
  ]
  node [
    id 4
    label "4"
  ]
  node [
    id 5
    label "5"
  ]
  node [
    id 6
    label "6"
  ]
  node [
    id 7
    label "7"
  ]
  node [
    id 8
    label "8"
  ]
  node [
    id 9
    label "9"
  ]
  node [
    id 10
    label "10"
  ]
  node [
    id 11
    label "11"
  ]
  node [
    id 12
    label "12"
  ]
  node [
    id 13
    label "13"
  ]
  edge [
    source 0
    target 1
    capacity 3099
  ]
  edge [
    source 0
    target 2
    capacity 4474
  ]
  edge [
    source 0
    target 7
    capacity 2722
  ]
  edge [
    source 1
    target 2
    capacity 3208
  ]
  edge [
    source 1
    target 3
    capacity 4312
  ]
  edge [
    source 1
    target 0
    capacity 3099
  ]
  edge [
    source 2
    target 5
    capacity 2884
  ]
  edge [
    source 2
    target 0
    capacity 4474
  ]
  edge [
    source 2
    target 1
    capacity 3208
  ]
  edge [
    source 3
    target 4
    capacity 4790
  ]
  edge [
    source 3
    target 10
    capacity 4514
  ]
  edge [
    source 3
    target 1
    capacity 4312
  ]
  edge [
    source 4
    target 5
    capacity 4437
  ]
  edge [
    source 4
    target 6
    capacity 1112
  ]
  edge [
    source 4
    target 3
    capacity 4790
  ]
  edge [
    source 5
    target 9
    capacity 3376
  ]
  edge [
    source 5
    target 12
    capacity 2921
  ]
  edge [
    source 5
    target 2
    capacity 2884
  ]
  edge [
    source 5
    target 4
    capacity 4437
  ]
  edge [
    source 6
    target 7
    capacity 3164
  ]
  edge [
    source 6
    target 4
    capacity 1112
  ]
  edge [
    source 7
    target 8
    capacity 4535
  ]
  edge [
    source 7
    target 0
    capacity 2722
  ]
  edge [
    source 7
    target 6
    capacity 3164
  ]
  edge [
    source 8
    target 9
    capacity 2673
  ]
  edge [
    source 8
    target 11
    capacity 1921
  ]
  edge [
    source 8
    target 13
    capacity 2983
  ]
  edge [
    source 8
    target 7
    capacity 4535
  ]
  edge [
    source 9
    target 5
    capacity 3376
  ]
  edge [
    source 9
    target 8
    capacity 2673
  ]
  edge [
    source 10
    target 11
    capacity 4710
  ]
  edge [
    source 10
    target 13
    capacity 1872
  ]
  edge [
    source 10
    target 3
    capacity 4514
  ]
  edge [
    source 11
    target 12
    capacity 3401
  ]
  edge [
    source 11
    target 8
    capacity 1921
  ]
  edge [
    source 11
    target 10
    capacity 4710
  ]
  edge [
    source 12
    target 13
    capacity 4321
  ]
  edge [
    source 12
    target 5
    capacity 2921
  ]
  edge [
    source 12
    target 11
    capacity 3401
  ]
  edge [
    source 13
    target 8
    capacity 2983
  ]
  edge [
    source 13
    target 10
    capacity 1872
  ]
  edge [
    source 13
    target 12
    capacity 4321
  ]
]
